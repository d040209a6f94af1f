"""``processdata/PrepareData_linear.py`` of the reference differs from ``PrepareData_linear_sing.py`` only in the percentile of
``loadBothConstraints`` (99.0 instead of 99.99, :88,97) -- the text-file leg that is out of scope here -- and in the default
``cell_No`` of GSE131811Module (22, :351).  The tile cut, the Splits/ files and the datasets are the same code."""
from .PrepareData_linear_sing import (GSE130711Module, gse131811Dataset, loadBothConstraints, splitPieces, stitchPieces,  # noqa: F401
                                      split_pieces_device, stitch_pieces_device, tile_origins)
from .PrepareData_linear_sing import GSE131811Module as _GSE131811Module


class GSE131811Module(_GSE131811Module):
    def __init__(self, batch_size=64, res=40000, piece_size=64, cell_line="Dros", cell_No=22, **kw):
        super().__init__(batch_size=batch_size, res=res, piece_size=piece_size, cell_line=cell_line, cell_No=cell_No, **kw)
