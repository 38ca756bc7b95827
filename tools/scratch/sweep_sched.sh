#!/bin/bash
# scheduler constants of the uniform round kernel (developer scratch): walk-steps/s and time-to-1spp per (trav_burst, wait_weight)
for bw in "3 8" "6 8" "10 8" "3 4" "10 4" "3 16" "2 8"; do set -- $bw; b=$1; w=$2
  python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --opt trav_burst=$b --opt wait_weight=$w 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('burst $b weight $w', round(r['value']/1e9,3), r['time_to_1spp_ms'])"
done
