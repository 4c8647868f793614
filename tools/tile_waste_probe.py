"""tile_waste_probe.py — CPU study (dev tool, uses the oracle's rulebooks): what fraction of the MFMAs issued by the
output-stationary conv kernels is useful, and how much a different ROW ORDER inside the tiles would recover.

A 16-row tile issues the MFMAs of offset k when ANY of its rows has that offset; rows that lack it multiply zeros.
useful = P / (16 * sum_tiles popcount(tile mask)).  Orders tried: canonical (b,z,y,x) rows; rows sorted by their
27-bit offset mask inside windows of W rows (W = 64 .. whole table)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tsm-det-pointcloud-_amd"))


def issued(mask_rows, tile=16):
    n = mask_rows.shape[0]
    pad = (-n) % tile
    m = np.concatenate([mask_rows, np.zeros(pad, mask_rows.dtype)]).reshape(-1, tile)
    t = np.bitwise_or.reduce(m, axis=1)
    return int(sum(bin(int(x)).count("1") for x in t)) * tile


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=int, default=2)
    ap.add_argument("--frames", type=int, default=4)
    args = ap.parse_args()
    from oracle import oracle as orc
    from pcdet_amd.datasets import synthetic as syn
    geom = syn.CONFIGS[args.cfg]["geom"]
    idx = []
    for b in range(args.frames):
        f = syn.make_frame(args.cfg, b)
        _v, c, _n = orc.voxelize(f["points"], geom["point_cloud_range"], geom["voxel_size"], 5, 400000)
        idx.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1))
    idx = np.concatenate(idx, 0)
    shape = [int(x) for x in (syn.grid_size_of(geom)[::-1] + [1, 0, 0])]
    levels = [("subm1", None)]
    geoms = [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1))]
    cur, cur_shape = idx, shape
    for lvl in range(4):
        pair = orc.subm_rulebook(cur, cur_shape)[0]
        n = cur.shape[0]
        has = (pair[:, :n] >= 0)
        P = int(has.sum())
        mask = np.zeros(n, np.int64)
        for k in range(has.shape[0]):
            mask |= has[k].astype(np.int64) << k
        base = issued(mask)
        line = "subm%d n=%6d P=%8d (%.1f/row)  canonical useful %.3f" % (lvl + 1, n, P, P / n, P / base)
        for W in (64, 256, 1024, 4096, n):
            order = np.arange(n)
            for s in range(0, n, W):
                seg = slice(s, min(n, s + W))
                order[seg] = s + np.argsort(mask[seg], kind="stable")
            line += " | W=%s %.3f" % ("all" if W == n else W, P / issued(mask[order]))
        # popcount-major then mask
        pc = np.array([bin(int(x)).count("1") for x in mask])
        order = np.lexsort((mask, pc))
        line += " | pc,mask all %.3f" % (P / issued(mask[order]))
        print(line, flush=True)
        if lvl < 3:
            k, s, p = geoms[lvl]
            out = orc.conv_rulebook(cur, cur_shape, k, s, p)
            cur = np.asarray(out[0])
            cur_shape = list(out[4])


if __name__ == "__main__":
    main()
