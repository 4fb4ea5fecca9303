# Round 6: bounded randomised differential stress at HEAD (tests/stress), the library as built
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06_stress
{
for seed in 51 7051 12051; do echo "## stress_raht.py $seed"; timeout 300 python tests/stress/stress_raht.py $seed 400 2>&1 | tail -1; done
echo "## stress_raht.py 22051, GPCC_REC=1"; GPCC_REC=1 timeout 300 python tests/stress/stress_raht.py 22051 400 2>&1 | tail -1
echo "## stress_raht.py 32051, GPCC_SWEEP=0"; GPCC_SWEEP=0 timeout 300 python tests/stress/stress_raht.py 32051 400 2>&1 | tail -1
echo "## stress_cx_batch.py 93"; timeout 600 python tests/stress/stress_cx_batch.py 93 2>&1 | tail -1
echo "## stress_lod.py 17"; timeout 600 python tests/stress/stress_lod.py 17 2>&1 | tail -1
echo "## stress_pred.py 23"; timeout 600 python tests/stress/stress_pred.py 23 2>&1 | tail -1
echo "## stress_raht_inter_gpu.py 29"; timeout 600 python tests/stress/stress_raht_inter_gpu.py 29 2>&1 | tail -1
} | tee gpurun_out/r06_stress/stress.txt
