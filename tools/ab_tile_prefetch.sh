# same-device A/B of three builds of the single-projection kernel: prefetch0 (the tile's first slices are requested after the
# prologue: rounds 1-3), prefetch1 (requested before the box / staging / column setup), rolling (prefetch1 + the next group of slices
# requested before the current one is updated). Libraries under tools/_ab/libs (make -C paris_amd/csrc EXTRA=-DPARIS_TILE_PREFETCH=0 / -DPARIS_TILE_ROLLING=1)
L=tools/_ab/libs
C="--cpu-budget 0 --cpu-c1 0 --live-traffic 0 --workloads 0 --paris-loop 0 --fused-steps 1"
cfgs=()
for w in "--steps 6 --warmup 1 --batch 48 --spread 1" "--workload c3 --slices 256 --steps 10 --warmup 2" "--workload c2 --steps 10 --warmup 2" "--workload c1 --steps 10 --warmup 2" "--workload c5 --steps 4 --warmup 1 --batch 36 --spread 1"; do
  for l in prefetch0 prefetch1 rolling; do cfgs+=("lib=$L/$l.so $w"); done
done
python tools/ab_args.py --rounds 3 --common "$C" "${cfgs[@]}"
