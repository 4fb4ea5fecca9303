"""Seeded clouds and parameter variants shared by the LoD golden generator
and the LoD tests (CPU oracle and GPU)."""
from mpeg_pcc_tmc13_amd import lod_params, synth

CLOUDS = ["rand5", "one", "two", "rand3k", "dups", "dense20k", "lidar15k", "sparse"]

VARIANTS = [dict(), dict(decimation=1), dict(decimation=2), dict(distribution=False), dict(dist2=1),
            dict(lifting=False, intra_range=64), dict(lifting=False, intra_range=64, blend=True),
            dict(bias=(1, 2, 1)), dict(inter_range=8), dict(neighbours=2), dict(levels=3),
            dict(decimation=1, sampling_period=2, levels=21), dict(decimation=2, sampling_period=3, dist2=1),
            dict(decimation=2, sampling_period=1)]


def make_cloud(name):
    return {
        "rand5": lambda: synth.random_cloud(5, seed=24, bits=2)[0],
        "one": lambda: synth.random_cloud(1, seed=1, bits=3)[0],
        "two": lambda: synth.random_cloud(2, seed=1, bits=3)[0],
        "rand3k": lambda: synth.random_cloud(3000, seed=2, bits=5)[0],
        "dups": lambda: synth.random_cloud(400, seed=9, bits=2, dup_fraction=0.3)[0],
        "dense20k": lambda: synth.dense_cloud(20000, seed=4, bits=8)[0],
        "lidar15k": lambda: synth.lidar_cloud(15000, seed=3)[0],
        "sparse": lambda: synth.random_cloud(3000, seed=8, bits=20)[0],
    }[name]()


def make_params(kw):
    lp = lod_params(**kw)
    if kw.get("lifting") is False:
        lp.intra_lod_prediction_skip_layers = 0
    return lp
