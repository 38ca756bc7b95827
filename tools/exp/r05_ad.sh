for sh in 0 1 2 3; do echo "== WOST3_G_SHIFT=$sh fused"; WOST3_G_FUSED=1 WOST3_G_SHIFT=$sh python tools/probes/bench3d_guided_only.py 2>/dev/null | tail -1; done
for sh in 0 1 2 3; do echo "== WOST3_G_SHIFT=$sh per depth"; WOST3_G_FUSED=0 WOST3_G_SHIFT=$sh python tools/probes/bench3d_guided_only.py 2>/dev/null | tail -1; done
