"""Operator-level wall time (what a tmc3 user sees as
"<attr>s processing time"): AttributeEncoder::encode + AttributeDecoder::decode of one
1 M-point slice through the reference's operator with the device inside
(oracle/_ref/libtmc3_shim3.so: host buffers, PCIe, entropy coder on the host, all
included) next to the unmodified build (libtmc3_ref.so) on one core.
    python tools/operator_time.py [n]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
CASES = {
    "raht_refl_lidar": dict(cloud="lidar", n=n, seed=5, transform=0, qp=34, chroma=0, subnode=1, search_range=2500),
    "raht_colour_dense": dict(cloud="dense", n=n, seed=4, transform=0, qp=34, chroma=-1, subnode=1, search_range=50000, bits=11),
    "lifting_colour_dense": dict(cloud="dense", n=n, seed=8, transform=2, qp=34, chroma=-1, subnode=1, search_range=50000, bits=11),
}
for name, case in CASES.items():
    row = {}
    for lib in ("libtmc3_ref.so", "libtmc3_shim3.so"):
        env = dict(os.environ, GPCC_STRICT="1")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "shim_operator_worker.py"),
                            json.dumps(dict(case, lib=lib, repeat=1))], capture_output=True, text=True, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        row[lib] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    a, b = row["libtmc3_ref.so"], row["libtmc3_shim3.so"]
    assert a["payload_md5"] == b["payload_md5"] and a["rec_dec_md5"] == b["rec_dec_md5"]
    print("%-22s n=%d  reference operator %.3f s   with the device inside %.3f s   (x%.1f)  payload %d B identical"
          % (name, n, a["seconds"], b["seconds"], a["seconds"] / b["seconds"], a["payload_len"]))
