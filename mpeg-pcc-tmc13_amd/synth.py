"""Seeded synthetic point clouds for parity tests and the benchmark
(SURVEY.md section 8(d)): S-dense (cat1-like voxelised surface with a smooth
colour field) and S-lidar (Ford_01_q_1mm-like spinning-lidar sweep with
reflectance).  Pure numpy; the same arrays feed the CPU oracle / reference
and the GPU path."""
import numpy as np


def morton_codes(xyz):
    """mortonAddr (reference tmc3/PCCMath.h:606-616): x -> bit 2, y -> bit 1,
    z -> bit 0 of every triplet.  numpy restatement for test-data setup."""
    def spread(v):
        x = v.astype(np.uint64) & np.uint64(0x1FFFFF)
        x = (x | (x << np.uint64(32))) & np.uint64(0x001F00000000FFFF)
        x = (x | (x << np.uint64(16))) & np.uint64(0x001F0000FF0000FF)
        x = (x | (x << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
        x = (x | (x << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
        x = (x | (x << np.uint64(2))) & np.uint64(0x1249249249249249)
        return x
    xyz = np.asarray(xyz)
    m = (spread(xyz[:, 0]) << np.uint64(2)) | (spread(xyz[:, 1]) << np.uint64(1)) | spread(xyz[:, 2])
    return m.astype(np.int64)


def sort_by_morton(xyz, attrs):
    """std::sort of MortonCodeWithIndex (code, then index): stable argsort."""
    codes = morton_codes(xyz)
    order = np.argsort(codes, kind="stable")
    return codes[order], np.ascontiguousarray(attrs[order]), order.astype(np.int32)


def _colour_field(p, rng, bitdepth, noise):
    """Smooth field + uniform noise, then a BT.709-like YCbCr mix so that the
    three components have the statistics the codec sees
    (convertPlyColourspace: 1)."""
    f = p / max(1.0, float(p.max()))
    r = 0.5 + 0.5 * np.sin(6.0 * f[:, 0] + 2.0 * f[:, 1])
    g = 0.5 + 0.5 * np.sin(5.0 * f[:, 1] + 3.0 * f[:, 2] + 1.0)
    b = 0.5 + 0.5 * np.cos(4.0 * f[:, 2] + 2.5 * f[:, 0])
    rgb = np.stack([r, g, b], 1) * ((1 << bitdepth) - 1)
    rgb += rng.integers(-noise, noise + 1, size=rgb.shape)
    rgb = np.clip(rgb, 0, (1 << bitdepth) - 1)
    half = 1 << (bitdepth - 1)
    y = 0.2126 * rgb[:, 0] + 0.7152 * rgb[:, 1] + 0.0722 * rgb[:, 2]
    cb = (rgb[:, 2] - y) / 1.8556 + half
    cr = (rgb[:, 0] - y) / 1.5748 + half
    out = np.stack([y, cb, cr], 1)
    return np.clip(np.rint(out), 0, (1 << bitdepth) - 1).astype(np.int32)


def dense_cloud(n, seed=1, bits=10, bitdepth=8, noise=12, dedup=True):
    """S-dense: points on a noisy sphere + torus, voxelised to `bits` bits.
    Returns (xyz int32 [m,3], colour int32 [m,3]) with m <= n after
    de-duplication (m == n if dedup=False, duplicates kept)."""
    rng = np.random.default_rng(seed)
    side = float((1 << bits) - 1)
    over = int(n * 1.35) + 64 if dedup else n
    k = over // 2
    u = rng.random(k) * 2 * np.pi
    v = np.arccos(1 - 2 * rng.random(k))
    rad = 0.36 + 0.004 * rng.standard_normal(k)
    sph = np.stack([rad * np.sin(v) * np.cos(u), rad * np.sin(v) * np.sin(u), rad * np.cos(v)], 1)
    k2 = over - k
    a = rng.random(k2) * 2 * np.pi
    b = rng.random(k2) * 2 * np.pi
    rr = 0.11 + 0.003 * rng.standard_normal(k2)
    tor = np.stack([(0.30 + rr * np.cos(b)) * np.cos(a), rr * np.sin(b) * 1.6, (0.30 + rr * np.cos(b)) * np.sin(a)], 1)
    p = np.concatenate([sph, tor], 0) + 0.5
    xyz = np.clip(np.rint(p * side), 0, side).astype(np.int32)
    if dedup:
        codes = morton_codes(xyz)
        _, first = np.unique(codes, return_index=True)
        first = np.sort(first)
        if len(first) > n:
            first = np.sort(rng.choice(first, size=n, replace=False))
        xyz = xyz[first]
    col = _colour_field(xyz.astype(np.float64), rng, bitdepth, noise)
    return np.ascontiguousarray(xyz), np.ascontiguousarray(col)


def lidar_cloud(n, seed=1, bits=18, rings=64, bitdepth=8, dedup=True, refl_noise=6):
    """S-lidar: `rings` laser rings swept over azimuth with range noise, on an
    18-bit grid (cfg/sequences-cat3.yaml geometry precision), 8-bit
    reflectance.  Returns (xyz int32 [m,3], refl int32 [m,1]).
    refl_noise: amplitude of the per-point noise on the smooth reflectance
    field (6: the headline frame, nearly every coefficient quantises to zero
    at qp 34; 48: a textured field, a few per cent of them survive)."""
    rng = np.random.default_rng(seed)
    side = float((1 << bits) - 1)
    over = int(n * 1.02) + 64 if dedup else n
    ring = rng.integers(0, rings, over)
    az = rng.random(over) * 2 * np.pi
    elev = np.deg2rad(-24.8 + 26.8 * ring / max(1, rings - 1))
    # piecewise "scene": ground plane + a few walls -> range as f(azimuth)
    ground = np.where(elev < -0.02, 1.8 / np.maximum(1e-3, -np.sin(elev)), 1e9)
    wall = 18.0 + 10.0 * np.sin(3 * az) + 4.0 * np.sin(11 * az + 1.0)
    rng_m = np.minimum(np.minimum(ground, wall), 120.0)
    rng_m = rng_m * (1 + 0.002 * rng.standard_normal(over))
    x = rng_m * np.cos(elev) * np.cos(az)
    y = rng_m * np.cos(elev) * np.sin(az)
    z = rng_m * np.sin(elev)
    p = np.stack([x, y, z], 1)
    p = (p + 125.0) / 250.0
    xyz = np.clip(np.rint(p * side), 0, side).astype(np.int32)
    refl = 40 + 60 * (np.sin(0.15 * rng_m) + 1) + 30 * np.cos(5 * az) + rng.integers(-refl_noise, refl_noise + 1, over)
    refl = np.clip(np.rint(refl), 0, (1 << bitdepth) - 1).astype(np.int32)
    if dedup:
        codes = morton_codes(xyz)
        _, first = np.unique(codes, return_index=True)
        first = np.sort(first)
        if len(first) > n:
            first = np.sort(rng.choice(first, size=n, replace=False))
        xyz, refl = xyz[first], refl[first]
    return np.ascontiguousarray(xyz), np.ascontiguousarray(refl.reshape(-1, 1))


def random_cloud(n, seed=1, bits=6, c=3, bitdepth=8, dup_fraction=0.0):
    """Uniform random voxels in a small cube (dense occupancy -> every
    neighbour pattern occurs), optional exact duplicates."""
    rng = np.random.default_rng(seed)
    xyz = rng.integers(0, 1 << bits, size=(n, 3)).astype(np.int32)
    if dup_fraction > 0 and n > 1:
        k = int(n * dup_fraction)
        src = rng.integers(0, n, k)
        dst = rng.integers(0, n, k)
        xyz[dst] = xyz[src]
    attrs = rng.integers(0, 1 << bitdepth, size=(n, c)).astype(np.int32)
    return xyz, attrs
