"""Row N2 of the coverage table (north_star: "Morton-sorted"): what would another particle order buy the LDS tiles?

Offline geometry study, no GPU: a C5-like lattice section (dp 0.002, DH 1, walls left out) is binned into the cells the
context uses (2h + skin wide, skin 0.28 h) and ordered (a) column-major by cell as the library does (cell id = cx*ncy + cy),
(b) in strips of S cell rows (column-major inside a strip), (c) along a Morton curve of the cells.  For tiles of 128
consecutive particles (one workgroup of the 2-lanes-per-particle kernels) it reports
  staging   = particles in the cells the tile touches and their 3x3 neighbours / 128 -- what a tile has to copy into LDS,
  ranges    = contiguous index ranges that neighbourhood consists of (the kernels stage THREE ranges per tile),
  in-tile   = fraction of the tile's neighbour pairs (r < 2h) whose partner lies in the tile itself (the pairs a
              both-partners-at-once evaluation could share).
python tools/tile_locality.py"""
import numpy as np
from scipy.spatial import cKDTree

dp, DH, DL = 0.002, 1.0, 0.6
h = 1.3 * dp
cs = 2 * h + 0.28 * h
x = np.arange(dp / 2, DL, dp)
y = np.arange(dp / 2, DH, dp)
X, Y = np.meshgrid(x, y, indexing="ij")
rng = np.random.default_rng(1)
pos = np.stack([X.ravel(), Y.ravel()], 1) + (rng.random((X.size, 2)) - 0.5) * 0.4 * dp
ncx, ncy = int(np.floor(DL / cs)), int(np.ceil(DH / cs)) + 1
cx = np.minimum((pos[:, 0] / (DL / ncx)).astype(int), ncx - 1)
cy = np.minimum((pos[:, 1] / cs).astype(int), ncy - 1)
tree = cKDTree(pos)
pairs = tree.query_pairs(2 * h, output_type="ndarray")


def morton(a, b):
    def spread(v):
        v = v.astype(np.uint64)
        v = (v | (v << 16)) & 0x0000FFFF0000FFFF
        v = (v | (v << 8)) & 0x00FF00FF00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0F
        v = (v | (v << 2)) & 0x3333333333333333
        v = (v | (v << 1)) & 0x5555555555555555
        return v
    return spread(a) | (spread(b) << np.uint64(1))


def study(name, key):
    order = np.argsort(key, kind="stable")
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    tile = rank // 128
    n_tiles = int(tile.max()) + 1
    # in-tile pair fraction
    same = tile[pairs[:, 0]] == tile[pairs[:, 1]]
    # staging: cells touched by each tile, dilated by one cell, particles therein; ranges: runs of consecutive ranks
    cell = cx * ncy + cy
    cnt = np.bincount(cell, minlength=ncx * ncy)
    cell_rank_lo = np.full(ncx * ncy, -1)
    first = np.zeros(ncx * ncy, dtype=np.int64)
    srt = np.sort(rank)  # noqa: F841
    lo = np.full(ncx * ncy, np.iinfo(np.int64).max)
    np.minimum.at(lo, cell, rank)
    staged, ranges = [], []
    sel = np.linspace(ncx * 0.25 * ncy * 9 // 128, n_tiles * 0.75, 60).astype(int)  # interior tiles
    for t in sel:
        cells = np.unique(cell[order[t * 128:(t + 1) * 128]])
        ccx, ccy = cells // ncy, cells % ncy
        nb = set()
        for ox in (-1, 0, 1):
            for oy in (-1, 0, 1):
                ok = (ccx + ox >= 0) & (ccx + ox < ncx) & (ccy + oy >= 0) & (ccy + oy < ncy)
                nb.update(((ccx + ox) * ncy + ccy + oy)[ok].tolist())
        nb = np.array(sorted(nb))
        staged.append(cnt[nb].sum() / 128.0)
        # ranges of ranks: a cell's particles are contiguous in every order studied; count the runs of adjacent cells
        los = np.sort(lo[nb][cnt[nb] > 0])
        his = los + cnt[nb][cnt[nb] > 0][np.argsort(lo[nb][cnt[nb] > 0])]
        ranges.append(1 + int(np.sum(los[1:] != his[:-1])))
    print(f"{name:34s} staging {np.mean(staged):5.2f} x   ranges {np.mean(ranges):5.1f}   in-tile pairs {100 * same.mean():5.1f} %")


print(f"{pos.shape[0]} particles, {ncx} x {ncy} cells of {cs / h:.2f} h, {pairs.shape[0] / pos.shape[0] * 2:.1f} neighbours per particle")
study("column-major cells (the library)", cx.astype(np.int64) * ncy + cy)
for S in (2, 4, 8):
    study(f"strips of {S} cell rows", (cy // S).astype(np.int64) * (ncx * S) + cx * S + cy % S)
study("Morton curve of the cells", morton(cx, cy).astype(np.int64))
