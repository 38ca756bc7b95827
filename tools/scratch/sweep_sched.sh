#!/bin/bash
# scheduler constants of the uniform round kernel (developer scratch): walk-steps/s and lane fill per (trav_burst, wait_weight)
for bw in "3 8" "6 8" "10 8" "6 4" "10 4" "16 4" "10 2" "5 6"; do set -- $bw; b=$1; w=$2
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-1spp --opt trav_burst=$b --opt wait_weight=$w 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); s = r['scheduler']; print('burst $b weight $w', round(r['value']/1e9,3), round(s['trav_lane_fill'],3), round(s['step_lane_fill'],3), round(s['trav_trips_per_wave_step'],2), round(s['step_trips_per_wave_step'],2))"
done
