#!/bin/bash
# In-pass probe of tile candidates: re-runs bench.py with the committed tune table in which the keys matching $1 (a
# prefix of the key line) are forced to each candidate in turn, and prints the step time and the per-shape lines matching $2.
#   bash tools/tune_variants_probe.sh "0 0 16384 1280 1280 1280 4 1 " "M=16384 N=1280 K=1280 bias res" 200 1 2 4 10 100
cd "$(dirname "$0")/.."
PREFIX="$1"; SHAPE="$2"; shift 2
mkdir -p gpurun_out/tune_var
for c in "$@"; do
  python3 - "$PREFIX" $c <<'P'
import sys
pre, c = sys.argv[1], sys.argv[2]
out = []
for l in open("profiles/r04c_tune_sdxl_1024_b2_r4.txt").read().splitlines():
    out.append(" ".join(l.split()[:-1]) + " " + c if l.startswith(pre) else l)
open("/tmp/tune_%s.txt" % c, "w").write("\n".join(out) + "\n")
P
  SMI_TUNE_FILE=/tmp/tune_$c.txt SMI_PROF_DUMP=1 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/tune_var/b$c.json 2> gpurun_out/tune_var/b$c.err
  echo "cand $c: $(python3 -c "import json;d=json.loads(open('gpurun_out/tune_var/b$c.json').read().strip().splitlines()[-1]);print(round(d['ms_per_step'],2))") ms/step"
  grep "$SHAPE" gpurun_out/tune_var/b$c.err | head -3
done
