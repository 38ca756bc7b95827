#!/bin/bash
# round 5, trip al: the tail of the launches per depth in launches of K depths with the survivors compacted in between (WOST3_G_TAIL_SEG),
# tools/probes/g3_forms_by_frame.py (8 samples, 4 trained; ms per solve)
for seg in 0 4 8 16; do echo "== WOST3_G_TAIL_SEG=$seg"; WOST3_G_TAIL_SEG=$seg G3_FRAMES="512 1024" G3_FORMS=0 python tools/probes/g3_forms_by_frame.py 2>&1 | grep -v amdgpu.ids; done
