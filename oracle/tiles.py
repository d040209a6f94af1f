"""CPU restatement of the reference's tile producer (TEST INFRASTRUCTURE: imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product path).

  tile_origins / split_pieces   processdata/PrepareData_linear_sing.py:25-46 (splitPieces: zero-pad the matrix to a
                                multiple of piece_size, walk i over rows and j >= i over columns in steps of `step`,
                                keep a tile when |i-j| <= int(piece_size*4*scal+1), scal = int(40000/resol), and
                                i+step, j+step stay inside the padded matrix)
  degrade                       :194-202 with the 'deno' H (src/functions/svd_replacement.py:148-168: H = H_pinv = identity
                                on the flattened tile): sample = tile.flat + sigma_0*z, noisy = sample reshaped
  stitch_pieces                 no reference counterpart (the reference never reassembles a matrix); the definition the
                                HIP kernel follows: element (r,c) <- the tile that holds it, else the tile that holds
                                (c,r), else 0.  Pinned by the round trip split -> stitch on symmetric matrices.
"""
import numpy as np


def tile_origins(n, piece_size, step, resol):
    scal = int(40000 / resol)
    rest = n % piece_size
    bound = n if rest == 0 else n + piece_size - rest
    band = int(piece_size * 4 * scal + 1)
    out = []
    for i in range(0, bound, step):
        for j in range(i, bound, step):
            if abs(i - j) <= band and i + step <= bound and j + step <= bound:
                out.append((i, j))
    return np.asarray(out, dtype=np.int64).reshape(-1, 2), bound


def split_pieces(mat, piece_size, step, resol):
    mat = np.asarray(mat)
    assert mat.shape[0] == mat.shape[1]
    org, bound = tile_origins(mat.shape[0], piece_size, step, resol)
    padded = np.zeros((bound, bound), dtype=mat.dtype)
    padded[:mat.shape[0], :mat.shape[1]] = mat
    pieces = [padded[i:i + piece_size, j:j + piece_size] for i, j in org]
    for p in pieces:
        if p.shape != (piece_size, piece_size):
            raise ValueError("inhomogeneous tile shapes (step < piece_size runs past the padded edge)")
    if not pieces:
        return np.zeros((0,), dtype=mat.dtype)[:, None]      # np.expand_dims(np.asarray([]), 1): shape (0, 1)
    return np.expand_dims(np.asarray(pieces), 1)


def degrade(tiles, sigma_0, z):
    t = np.asarray(tiles, dtype=np.float32)
    sample = t.reshape(t.shape[0], -1) + np.float32(sigma_0) * np.asarray(z, dtype=np.float32).reshape(t.shape[0], -1)
    return sample.reshape(t.shape), sample


def stitch_pieces(tiles, origins, n):
    tiles = np.asarray(tiles)
    if tiles.ndim == 4:
        tiles = tiles[:, 0]
    out = np.zeros((n, n), dtype=tiles.dtype)
    have = np.zeros((n, n), dtype=bool)
    p = tiles.shape[-1] if len(tiles) else 0
    for t, (i, j) in enumerate(origins):
        h, w = max(0, min(p, n - i)), max(0, min(p, n - j))
        out[i:i + h, j:j + w] = tiles[t, :h, :w]
        have[i:i + h, j:j + w] = True
    fill = ~have & have.T
    out[fill] = out.T[fill]
    return out
