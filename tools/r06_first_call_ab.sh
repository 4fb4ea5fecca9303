# Round 6: the first_call leg (fresh context + gpcc_ctx_reserve + one forward transform, 20 trials) three times, with the
# reserve writing the memory it allocates (default) and without (GPCC_RESERVE_TOUCH=0)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06_first_call
for v in 1 0; do for k in 1 2 3; do
  GPCC_RESERVE_TOUCH=$v python bench.py --no-extras --steps 3 --no-cpu-baseline --no-profile --legs first_call > gpurun_out/r06_first_call/t${v}_$k.json 2>/dev/null
done; done
python - <<PY
import json
for v in (1,0):
    for k in (1,2,3):
        d=json.loads(open("gpurun_out/r06_first_call/t%d_%d.json"%(v,k)).read().splitlines()[0])["first_call"]
        print("touch",v,"run",k,"reserved",{a:b for a,b in d["reserved"].items() if a!="trials"},"| unreserved",{a:b for a,b in d["unreserved"].items() if a!="trials"})
PY
