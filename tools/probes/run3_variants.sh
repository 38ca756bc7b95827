# developer probe: the 3-D bench scenes under the knobs of the wave-cooperative queries, one spec per line
while read -r spec; do
  [ -z "$spec" ] && continue
  echo "== $spec"; env $spec python tools/probes/bench3d_only.py 2 2>&1 | grep -v amdgpu.ids | tail -1
done <<LIST
WOST3_POOL_CAP=512
WOST3_POOL_CAP=384
WOST3_POOL_CAP=768
WOST3_POOL_CAP=512 WOST3_CP_TRIGGER=32
WOST3_WAVE=0 WOST3_COOP=0
LIST
