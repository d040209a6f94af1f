"""Tile producer / stitcher counterpart of the reference's ``processdata`` package."""
from .PrepareData_linear_sing import (GSE130711Module, GSE131811Module, gse131811Dataset, splitPieces, stitchPieces,  # noqa: F401
                                      split_pieces_device, stitch_pieces_device, tile_origins)
