# compare prebuilt variants of libwost_hip.so (elaina_amd/lib/variants/*.so, e.g. other compiler flags) on config 2
cp elaina_amd/lib/libwost_hip.so /tmp/libwost_hip.keep
for f in elaina_amd/lib/variants/*.so; do
  cp "$f" elaina_amd/lib/libwost_hip.so
  for i in 1 2; do
    python bench.py --config 2 --no-extras --no-cpu-baseline --no-1spp 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'])"
  done
done
cp /tmp/libwost_hip.keep elaina_amd/lib/libwost_hip.so
