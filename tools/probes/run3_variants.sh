# developer probe: the 3-D bench scenes under the knobs of the wave-cooperative queries (argument: a file of one spec per line, or the default list)
while read -r spec; do
  [ -z "$spec" ] && continue
  echo "== $spec"; env $spec python tools/probes/bench3d_only.py 2 2>&1 | grep -v amdgpu.ids | tail -1
done <<LIST
WOST3_POOL_CAP=512
WOST3_POOL_CAP=640
WOST3_POOL_CAP=768
WOST3_POOL_CAP=1024 WOST3_BLOCKS_PER_CU=2
WOST3_POOL_CAP=512 WOST3_CP_TRIGGER=32
WOST3_POOL_CAP=512 WOST3_CP_TRIGGER=16
WOST3_POOL_CAP=512 WOST3_CP_TRIGGER=8
WOST3_POOL_CAP=512 WOST3_RAY_TRIGGER=16
WOST3_POOL_CAP=512 WOST3_RAY_TRIGGER=8
WOST3_POOL_CAP=512 WOST3_BLOCKS_PER_CU=3
WOST3_POOL_CAP=512 WOST3_BLOCKS_PER_CU=2
LIST
