"""Deterministic synthetic KITTI- / Waymo-shaped point clouds.

Real datasets are absent (no network); the reference's loaders (pcdet/datasets/kitti, waymo) are out of
scope.  The generator draws the ACTIVE-VOXEL SET first so counts are pinned, then points inside those
voxels:
  1. ground sheet (70 %): scan arcs around the sensor whose radii grow geometrically (areal density ~ 1/r, returns
     along a ring are voxel-contiguous like real scan lines), z index = plane + {0,1};
  2. object / wall sheets (30 %): surface voxels of axis-aligned and 30-degree-rotated boxes
     (4 x 1.8 x 1.6 m "cars") and 20 x 3 m walls; de-duplicated to exactly `n_active` voxels;
  3. per voxel 1 + Poisson(lambda) points uniform inside the cell (clipped to `n_points` total),
     intensity ~ U[0,1] (Waymo: elongation ~ U[0,1.5]); shuffled.
Seeds: numpy.random.default_rng(1000 * cfg_id + frame_idx).
"""
import numpy as np

#: dataset geometry — reference tools/cfgs/dataset_configs/kitti_dataset.yaml:4,80-86 and
#: waymo_dataset.yaml:6,73-79
KITTI = dict(name="kitti", point_cloud_range=[0.0, -40.0, -3.0, 70.4, 40.0, 1.0], voxel_size=[0.05, 0.05, 0.1],
             num_point_features=4, max_points_per_voxel=5, max_voxels=dict(train=16000, test=40000),
             sensor_xy=(0.0, 0.0), class_names=["Car", "Pedestrian", "Cyclist"])
WAYMO = dict(name="waymo", point_cloud_range=[-75.2, -75.2, -2.0, 75.2, 75.2, 4.0], voxel_size=[0.1, 0.1, 0.15],
             num_point_features=5, max_points_per_voxel=5, max_voxels=dict(train=150000, test=150000),
             sensor_xy=(0.0, 0.0), class_names=["Vehicle", "Pedestrian", "Cyclist"])

#: reduced KITTI-like geometry for fast CPU / parity tests (same voxel size, 12.8 m x 16 m range -> BEV 32 x 40)
MINI = dict(name="kitti", point_cloud_range=[0.0, -8.0, -3.0, 12.8, 8.0, 1.0], voxel_size=[0.05, 0.05, 0.1],
            num_point_features=4, max_points_per_voxel=5, max_voxels=dict(train=16000, test=40000),
            sensor_xy=(0.0, 0.0), class_names=["Car", "Pedestrian", "Cyclist"])

#: BASELINE.json configs[0..4] -> (geometry, points/frame, active voxels/frame, batch)
CONFIGS = {
    0: dict(geom=MINI, n_points=4000, n_active=1500, batch=2, n_obj=4),
    1: dict(geom=KITTI, n_points=16384, n_active=5000, batch=1, crop=True),
    2: dict(geom=KITTI, n_points=20000, n_active=16000, batch=4),
    3: dict(geom=WAYMO, n_points=180000, n_active=80000, batch=2),
    4: dict(geom=WAYMO, n_points=180000, n_active=80000, batch=2),
    5: dict(geom=WAYMO, n_points=600000, n_active=300000, batch=1, max_voxels=400000),
    # not a BASELINE config: the Waymo geometry (full 1504 x 1504 x 40 grid, C = 5, Waymo yaml) at a voxel count the CPU
    # oracle affords, for detector-level parity tests of the Waymo model
    6: dict(geom=WAYMO, n_points=18000, n_active=8000, batch=2, n_obj=12),
}


def grid_size_of(geom):
    r = np.asarray(geom["point_cloud_range"], np.float64)
    v = np.asarray(geom["voxel_size"], np.float64)
    return np.round((r[3:6] - r[0:3]) / v).astype(np.int64)  # (gx, gy, gz); data_processor.py:129-130


def _box_surface(rng, cx, cy, z0, lx, ly, lz, yaw, vs, n):
    """n random points on the vertical faces + top of an oriented box."""
    u = rng.random(n)
    face = rng.integers(0, 5, n)
    a = (rng.random(n) - 0.5)
    b = (rng.random(n) - 0.5)
    x = np.where(face < 2, (face * 2 - 1) * 0.5 * lx, a * lx)
    y = np.where(face < 2, b * ly, np.where(face < 4, ((face - 2) * 2 - 1) * 0.5 * ly, b * ly))
    z = np.where(face < 4, u * lz, lz)
    c, s = np.cos(yaw), np.sin(yaw)
    return np.stack([cx + c * x - s * y, cy + s * x + c * y, z0 + z], 1)


def make_frame(cfg_id, frame_idx=0, with_boxes=True):
    """Returns dict(points[Np, C] float32, gt_boxes[M, 8] float32 (x,y,z,dx,dy,dz,yaw,class 1..3))."""
    cfg = CONFIGS[cfg_id]
    geom = cfg["geom"]
    rng = np.random.default_rng(1000 * cfg_id + frame_idx)
    pr = np.asarray(geom["point_cloud_range"], np.float64)
    vs = np.asarray(geom["voxel_size"], np.float64)
    gs = grid_size_of(geom)
    n_active, n_points = cfg["n_active"], cfg["n_points"]
    lo, hi = pr[:3].copy(), pr[3:].copy()
    if cfg.get("crop"):
        # KITTI crop: a 35 x 40 m window in front of the sensor
        hi[0] = lo[0] + 35.2
        lo[1], hi[1] = -20.0, 20.0
    sx, sy = geom["sensor_xy"]

    keys = np.empty((0,), np.int64)
    cand_xyz = []
    boxes = []
    n_obj = cfg.get("n_obj", 10 if geom["name"] == "kitti" else 40)
    # objects / walls first so they always survive de-duplication
    n_obj_vox = int(0.3 * n_active)
    per = max(n_obj_vox // (n_obj + 4), 8)
    ground_z = lo[2] + 0.45 * (hi[2] - lo[2]) if geom["name"] == "kitti" else lo[2] + 0.3 * (hi[2] - lo[2])
    ground_z = float(np.floor((ground_z - pr[2]) / vs[2]) * vs[2] + pr[2])
    for j in range(n_obj):
        mrg = min(5.0, 0.25 * (hi[0] - lo[0]), 0.25 * (hi[1] - lo[1]))
        cx = rng.uniform(lo[0] + mrg, hi[0] - mrg)
        cy = rng.uniform(lo[1] + mrg, hi[1] - mrg)
        yaw = 0.0 if j % 2 == 0 else np.deg2rad(30.0)
        cls = 1 + (j % 3)
        dims = [(4.0, 1.8, 1.6), (0.8, 0.6, 1.73), (1.76, 0.6, 1.73)][cls - 1]
        cand_xyz.append(_box_surface(rng, cx, cy, ground_z, dims[0], dims[1], dims[2], yaw, vs, per * 6))
        boxes.append([cx, cy, ground_z + dims[2] / 2, dims[0], dims[1], dims[2], yaw, cls])
    for j in range(4):
        mrg = min(12.0, 0.4 * (hi[0] - lo[0]), 0.4 * (hi[1] - lo[1]))
        cx = rng.uniform(lo[0] + mrg, hi[0] - mrg)
        cy = rng.uniform(lo[1] + mrg, hi[1] - mrg)
        cand_xyz.append(_box_surface(rng, cx, cy, ground_z, 20.0, 0.2, 3.0, rng.uniform(0, np.pi), vs, per * 12))
    obj = np.concatenate(cand_xyz, 0)

    def to_keys(xyz):
        c = np.floor((xyz - pr[:3]) / vs).astype(np.int64)
        ok = np.all((c >= 0) & (c < gs), 1) & np.all((xyz >= lo) & (xyz < hi), 1)
        c = c[ok]
        return (c[:, 2] * gs[1] + c[:, 1]) * gs[0] + c[:, 0]

    k_obj = np.unique(to_keys(obj))
    rng.shuffle(k_obj)
    k_obj = k_obj[:n_obj_vox]
    keys = k_obj
    # ground sheet: LiDAR-like scan ARCS.  Ring radii grow geometrically (r_i = r0 * gamma^i), every ring is a
    # voxel-contiguous arc, so the areal voxel density falls off ~ 1/r as the spec asks while neighbouring returns
    # stay adjacent the way real scan lines do (isolated random voxels would dilate 8x at every strided conv).
    rmax = float(np.hypot(max(abs(lo[0] - sx), abs(hi[0] - sx)), max(abs(lo[1] - sy), abs(hi[1] - sy))))
    r0 = 2.5
    need_ground = n_active - keys.size

    def ring_keys(radii, frac):
        out = []
        for r in radii:
            dth = 0.45 * min(vs[0], vs[1]) / r
            th0 = rng.random() * 2 * np.pi
            th = th0 + np.arange(0.0, 2 * np.pi * frac, dth)
            rr = r + 0.35 * vs[0] * np.sin(7.0 * th + rng.random() * 6.28)        # mild radial wobble
            x, y = sx + rr * np.cos(th), sy + rr * np.sin(th)
            zc = np.floor(th * (3.0 + 40.0 / r)).astype(np.int64) % 2             # z index plane + {0,1} in runs
            z = ground_z + (zc - 0.5) * vs[2] * 0.9 + 0.5 * vs[2]
            out.append(to_keys(np.stack([x, y, z], 1)))
        k = np.unique(np.concatenate(out)) if out else np.empty((0,), np.int64)
        return np.setdiff1d(k, keys, assume_unique=False)

    n_rings = 8
    while True:
        gamma = (rmax / r0) ** (1.0 / n_rings)
        radii = r0 * gamma ** (np.arange(n_rings) + 0.5)
        k_full = ring_keys(radii, 1.0)
        if k_full.size >= need_ground or n_rings > 4096:
            break
        n_rings = int(n_rings * 1.5) + 1
    frac = min(1.0, 1.02 * need_ground / max(k_full.size, 1))
    k_ground = ring_keys(radii, frac) if frac < 1.0 else k_full
    if k_ground.size < need_ground:          # top up from the full rings
        extra = np.setdiff1d(k_full, k_ground)
        rng.shuffle(extra)
        k_ground = np.concatenate([k_ground, extra[:need_ground - k_ground.size]])
    if k_ground.size > need_ground:          # trim the tail of the sorted key list (a contiguous far slab)
        k_ground = np.sort(k_ground)[:need_ground]
    keys = np.concatenate([keys, k_ground])
    while keys.size < n_active:              # pathological geometry only: random fill
        m = 4 * (n_active - keys.size) + 1024
        xyz = np.stack([lo[0] + rng.random(m) * (hi[0] - lo[0]), lo[1] + rng.random(m) * (hi[1] - lo[1]),
                        np.full(m, ground_z + 0.5 * vs[2])], 1)
        k = np.setdiff1d(np.unique(to_keys(xyz)), keys)
        rng.shuffle(k)
        keys = np.concatenate([keys, k[:n_active - keys.size]])
    keys = keys[:n_active]

    # points: 1 + Poisson(lambda) per voxel, clipped to n_points
    lam = max(n_points / n_active - 1.0, 0.0)
    cnt = 1 + rng.poisson(lam, n_active)
    over = int(cnt.sum()) - n_points
    if over > 0:
        order = np.argsort(-cnt, kind="stable")
        i = 0
        while over > 0:
            j = order[i % n_active]
            if cnt[j] > 1:
                cnt[j] -= 1
                over -= 1
            i += 1
    vx = keys % gs[0]
    vy = (keys // gs[0]) % gs[1]
    vz = keys // (gs[0] * gs[1])
    rep = np.repeat(np.arange(n_active), cnt)
    # keep points strictly inside their cell (margin avoids fp32 boundary flips)
    u = 0.05 + 0.9 * rng.random((rep.size, 3))
    xyz = (np.stack([vx[rep], vy[rep], vz[rep]], 1) + u) * vs + pr[:3]
    feats = [rng.random(rep.size)]
    if geom["num_point_features"] == 5:
        feats.append(rng.random(rep.size) * 1.5)
    pts = np.concatenate([xyz, np.stack(feats, 1)], 1).astype(np.float32)
    rng.shuffle(pts, axis=0)
    out = dict(points=pts, n_active=int(n_active))
    if with_boxes:
        out["gt_boxes"] = np.asarray(boxes, np.float32)
    return out


def make_batch(cfg_id, batch=None, start_frame=0):
    """collate_batch-style dict (pcdet/datasets/dataset.py:161-229): points [N, 1+C] with a leading
    batch-index column, gt_boxes [B, Mmax, 8], batch_size."""
    cfg = CONFIGS[cfg_id]
    batch = cfg["batch"] if batch is None else batch
    frames = [make_frame(cfg_id, start_frame + i) for i in range(batch)]
    pts = [np.concatenate([np.full((f["points"].shape[0], 1), i, np.float32), f["points"]], 1)
           for i, f in enumerate(frames)]
    mmax = max(f["gt_boxes"].shape[0] for f in frames)
    gt = np.zeros((batch, mmax, 8), np.float32)
    for i, f in enumerate(frames):
        gt[i, :f["gt_boxes"].shape[0]] = f["gt_boxes"]
    return dict(points=np.concatenate(pts, 0), gt_boxes=gt, batch_size=batch,
                frame_id=np.arange(start_frame, start_frame + batch))
