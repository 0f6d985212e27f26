# frames per step x HIP streams sweep of bench.py (config B), inside one call so that the box is the same
for cfg in "32 2" "24 2" "48 2" "40 2" "36 3" "32 2" "48 2"; do set -- $cfg; echo -n "batch=$1 streams=$2: "; timeout 250 python bench.py --no-extras --no-cpu-baseline --batch $1 --streams $2 --steps $((1920 / $1)) 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
