"""Deterministic synthetic stand-ins for the reference's missing .raw assets.

The reference ships no volume data (`.MISSING_LARGE_BLOBS:1-4` lists
`assets/bonsai_256x256x256_uint8.raw`, `assets/boston_teapot_256x256x178_uint8.raw`
and `..._segments.raw`), so benchmarks and tests use these integer-only
generators.  All arithmetic is uint32/int64 with explicit
wrap-around, so the bytes are identical on every numpy build; tests pin the
SHA-256 of each output.  Files are produced "as on disk" (x fastest, then y,
then z); the loader applies the reference's pad/truncate + Y flip.
"""
import hashlib

import numpy as np

DEFAULT_SEED = 20250310

# the segment table the reference ships: assets/boston_teapot_256x256x178_uint8_segments.json
TEAPOT_SEGMENTS = [
    {"id": "Segment_4", "importance": 0, "index": 1, "label_value": 3, "name": "Cup"},
    {"id": "Segment_5", "importance": 0, "index": 2, "label_value": 4, "name": "Ground"},
    {"id": "Segment_2", "importance": 255, "index": 0, "label_value": 2, "name": "Lobster"},
]


def lowbias32(x):
    """Chris Wellons' lowbias32 integer hash on uint32 arrays (wrapping)."""
    x = np.asarray(x, dtype=np.uint32).copy()
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def _hash_scalar(v):
    with np.errstate(over="ignore"):
        return int(lowbias32(np.array([v & 0xFFFFFFFF], dtype=np.uint32))[0])


def _slice_hash(n_x, n_y, z, seed):
    with np.errstate(over="ignore"):
        x = np.arange(n_x, dtype=np.uint32)[None, :]
        y = np.arange(n_y, dtype=np.uint32)[:, None]
        idx = x + np.uint32(n_x) * (y + np.uint32(n_y) * np.uint32(z)) + np.uint32(seed & 0xFFFFFFFF)
        return lowbias32(idx)


def _bonsai_spheres(n, seed):
    """12 canopy spheres: centre in x,z in [.25,.75)n, y in [.52,.84)n, radius in [.10,.19)n."""
    out = []
    for i in range(12):
        h0 = _hash_scalar(seed * 31 + 4 * i + 0)
        h1 = _hash_scalar(seed * 31 + 4 * i + 1)
        h2 = _hash_scalar(seed * 31 + 4 * i + 2)
        h3 = _hash_scalar(seed * 31 + 4 * i + 3)
        cx = n // 4 + (h0 % 1024) * (n // 2) // 1024
        cy = (52 * n) // 100 + (h1 % 1024) * ((32 * n) // 100) // 1024
        cz = n // 4 + (h2 % 1024) * (n // 2) // 1024
        r = (10 * n) // 100 + (h3 % 1024) * ((9 * n) // 100) // 1024
        out.append((cx, cy, cz, max(r, 1)))
    return out


def synth_bonsai(n=256, seed=DEFAULT_SEED, with_labels=False):
    """n^3 uint8 'bonsai': pot (230, label 4), trunk (192..207, label 3), canopy of 12
    radial-falloff spheres (82..167, label 2), air noise 0..7 (label 0).  Roughly 10 % of
    the voxels are >= 39 (density_threshold 0.15)."""
    vol = np.empty((n, n, n), np.uint8)           # [z, y, x]
    lab = np.zeros((n, n, n), np.uint8) if with_labels else None
    spheres = _bonsai_spheres(n, seed)
    x = np.arange(n, dtype=np.int64)[None, :]
    y = np.arange(n, dtype=np.int64)[:, None]
    half = n // 2
    pot_y0, pot_y1 = (5 * n) // 100, (18 * n) // 100
    pot_h = (22 * n) // 100
    trunk_y1 = (55 * n) // 100
    trunk_r2 = ((45 * n) // 1000) ** 2
    for z in range(n):
        h = _slice_hash(n, n, z, seed)
        s = (h & np.uint32(7)).astype(np.int64)
        l = np.zeros((n, n), np.uint8) if with_labels else None
        # canopy (max over spheres)
        for (cx, cy, cz, r) in spheres:
            dz = z - cz
            if abs(dz) >= r:
                continue
            rr = int(np.sqrt(r * r - dz * dz)) + 1
            x0, x1 = max(cx - rr, 0), min(cx + rr + 1, n)
            y0, y1 = max(cy - rr, 0), min(cy + rr + 1, n)
            xs = x[:, x0:x1] - cx
            ys = y[y0:y1, :] - cy
            d2 = xs * xs + ys * ys + dz * dz
            inside = d2 < r * r
            noise = ((h[y0:y1, x0:x1] >> np.uint32(12)) & np.uint32(15)).astype(np.int64) - 8
            val = 160 - (70 * d2) // (r * r) + noise
            sub = s[y0:y1, x0:x1]
            upd = inside & (val > sub)
            sub[upd] = val[upd]
            if with_labels:
                l[y0:y1, x0:x1][upd] = 2
        # trunk
        dzt = z - half
        tr = ((x - half) ** 2 + dzt * dzt < trunk_r2) & (y >= pot_y1) & (y < trunk_y1)
        tv = 192 + ((h >> np.uint32(8)) & np.uint32(15)).astype(np.int64)
        s[tr] = tv[tr]
        # pot
        pot = (y >= pot_y0) & (y < pot_y1) & (np.abs(x - half) < pot_h) & (abs(dzt) < pot_h)
        s[pot] = 230
        if with_labels:
            l[tr] = 3
            l[pot] = 4
            lab[z] = l
        vol[z] = s.astype(np.uint8)
    return (vol.reshape(-1), lab.reshape(-1)) if with_labels else vol.reshape(-1)


def synth_teapot(nx=256, ny=256, nz=178, seed=DEFAULT_SEED):
    """Stand-in for boston_teapot_256x256x178: returns (density, labels), nx*ny*nz bytes each.
    Ground slab (label 4, ~70), cup shell (label 3, ~110), lobster blob inside (label 2, ~200);
    the labels are the label_values of the shipped segments JSON."""
    vol = np.empty((nz, ny, nx), np.uint8)
    lab = np.zeros((nz, ny, nx), np.uint8)
    x = np.arange(nx, dtype=np.int64)[None, :]
    y = np.arange(ny, dtype=np.int64)[:, None]
    cx, cy, cz = nx // 2, (47 * ny) // 100, nz // 2
    ro, ri = (27 * nx) // 100, (23 * nx) // 100
    g0, g1 = (10 * ny) // 100, (16 * ny) // 100
    lx, ly, lz = (13 * nx) // 100, (8 * ny) // 100, (11 * nx) // 100
    lcy = (42 * ny) // 100
    for z in range(nz):
        h = _slice_hash(nx, ny, z, seed + 1)
        s = (h & np.uint32(7)).astype(np.int64)
        l = np.zeros((ny, nx), np.uint8)
        dz = z - cz
        d2 = (x - cx) ** 2 + (y - cy) ** 2 + dz * dz
        shell = (d2 < ro * ro) & (d2 >= ri * ri) & (y >= g1) & (y < cy + (15 * ny) // 100)
        sv = 104 + ((h >> np.uint32(8)) & np.uint32(15)).astype(np.int64)
        s[shell] = sv[shell]
        l[shell] = 3
        # ellipsoid: (dx/lx)^2 + (dy/ly)^2 + (dz/lz)^2 < 1, in integers
        e = ((x - cx) ** 2) * (ly * ly * lz * lz) + ((y - lcy) ** 2) * (lx * lx * lz * lz) \
            + (dz * dz) * (lx * lx * ly * ly)
        blob = e < (lx * lx * ly * ly * lz * lz)
        bv = 192 + ((h >> np.uint32(16)) & np.uint32(15)).astype(np.int64)
        s[blob] = bv[blob]
        l[blob] = 2
        ground = (y >= g0) & (y < g1) & np.ones((1, nx), bool)
        gv = 66 + ((h >> np.uint32(20)) & np.uint32(7)).astype(np.int64)
        s[ground] = gv[ground]
        l[ground] = 4
        vol[z] = s.astype(np.uint8)
        lab[z] = l
    return vol.reshape(-1), lab.reshape(-1)


def synth_ball(n=256, seed=DEFAULT_SEED):
    """n^3 uint8: one ball of radius 0.36 n whose value rises from 48 at its surface to 208 at its centre (+- 8 noise): every
    ray through it is one long run of dense samples of slowly growing opacity.  A scene for the scheduling rows (long chains
    of dependent samples everywhere, no thin structure)."""
    vol = np.empty((n, n, n), np.uint8)
    x = np.arange(n, dtype=np.int64)[None, :]
    y = np.arange(n, dtype=np.int64)[:, None]
    c, r = n // 2, (36 * n) // 100
    for z in range(n):
        h = _slice_hash(n, n, z, seed + 7)
        s = (h & np.uint32(7)).astype(np.int64)
        d2 = (x - c) ** 2 + (y - c) ** 2 + (z - c) ** 2
        inside = d2 < r * r
        noise = ((h >> np.uint32(12)) & np.uint32(15)).astype(np.int64) - 8
        val = 208 - (160 * d2) // (r * r) + noise
        s[inside] = val[inside]
        vol[z] = np.clip(s, 0, 255).astype(np.uint8)
    return vol.reshape(-1)


def synth_vessels(n=256, seed=DEFAULT_SEED):
    """n^3 uint8: 48 thin tubes (radius 0.008-0.02 n, value 176..223) along slowly bending paths through the volume, air noise
    0..7: a sparse scene -- most rays that come near anything miss it, the hits are short dense runs."""
    vol = np.empty((n, n, n), np.uint8)
    x = np.arange(n, dtype=np.int64)[None, :]
    y = np.arange(n, dtype=np.int64)[:, None]
    tubes = []
    for i in range(48):
        h = [_hash_scalar(seed * 57 + 8 * i + k) for k in range(8)]
        # centre line (x, y) as a function of z: a + b * z / n + c * tri(z), integers in 1/1024 voxel units
        ax, ay = (n // 8 + (h[0] % 1024) * (3 * n // 4) // 1024) * 1024, (n // 8 + (h[1] % 1024) * (3 * n // 4) // 1024) * 1024
        bx, by = (h[2] % 2048 - 1024) * (n // 4), (h[3] % 2048 - 1024) * (n // 4)          # drift over the whole depth, +- n/4 voxels
        cx, cy = (h[4] % 1024) * (n // 16), (h[5] % 1024) * (n // 16)                      # wobble amplitude, up to n/16 voxels
        period = n // 4 + (h[6] % 1024) * (n // 2) // 1024
        rad = max((8 * n) // 1000 + (h[7] % 1024) * ((12 * n) // 1000) // 1024, 1)
        tubes.append((ax, ay, bx, by, cx, cy, max(period, 4), rad))
    for z in range(n):
        h = _slice_hash(n, n, z, seed + 11)
        s = (h & np.uint32(7)).astype(np.int64)
        tv = 176 + ((h >> np.uint32(8)) & np.uint32(47)).astype(np.int64)
        for (ax, ay, bx, by, cx, cy, period, rad) in tubes:
            ph = z % period
            tri = (4096 * ph) // period - 1024 if ph * 2 < period else 3072 - (4096 * ph) // period      # triangle wave in [-1024, 1024]
            px = (ax + (bx * z) // n + (cx * tri) // 1024) // 1024
            py = (ay + (by * z) // n + (cy * tri) // 1024) // 1024
            x0, x1 = max(px - rad, 0), min(px + rad + 1, n)
            y0, y1 = max(py - rad, 0), min(py + rad + 1, n)
            if x0 >= x1 or y0 >= y1:
                continue
            d2 = (x[:, x0:x1] - px) ** 2 + (y[y0:y1, :] - py) ** 2
            sub = s[y0:y1, x0:x1]
            upd = d2 <= rad * rad
            sub[upd] = tv[y0:y1, x0:x1][upd]
        vol[z] = s.astype(np.uint8)
    return vol.reshape(-1)


def constant_cube(n, value):
    return np.full(n * n * n, value, np.uint8)


def sha256(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
