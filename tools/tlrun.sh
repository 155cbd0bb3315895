# Tuning only: two-stamp timelines (row start + point k) of the fused / plain sweep, side builds libdfe_tlp<k>.so
for what in ${WHAT:-fused}; do for k in 2 3 4 5 6; do m=$((1 + (1<<k))); echo "== $what point $k"; DFE_LIB=tools/ubench/libdfe_tlp$k.so TL_MASK=$m timeout -k 10 100 python tools/timeline.py vga $what 2>&1 | grep -v amdgpu.ids | grep -v "block 100"; done; done
