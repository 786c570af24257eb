# PARIS's per-projection loop through paris::hip over shapes other than the natural volume (slabs, coarse and small volumes, ROI): is any
# build variant notably faster than the default anywhere -- i.e. does the default stall? Columns: loop GVox/s | host fill s | calls s | wait s
D=paris_amd/host/demo
run() { lbl=$1; shift; for v in demo demo_filter_deferral demo_serial; do echo -n "$lbl | $v: "; $D/paris_hip_$v "$@" lcg /dev/null --cycle 48 --no-out "${EXTRA[@]}" | sed -n 2,3p | tr "\n" " " | sed -e 's/projection loops \([0-9.]*\) s: \([0-9.]*\) GVox.*buffer) \([0-9.]*\) s, backend calls \([0-9.]*\) s, final wait for the GPU \([0-9.]*\) s.*/\2 GVox\/s | loop \1 s | fill \3 | calls \4 | wait \5/'; echo; done; }
EXTRA=(--vol 1024 1024 128 0.0994877); run "1024^2 x 1440 -> 1024x1024x128 slab" 1024 1024 0.2 0.2 0 0 500 500 0.25 1440
EXTRA=(--vol 512 512 512 0.3918664); run "2048^2 x 720 -> 512^3 coarse" 2048 2048 0.2 0.2 0 0 500 500 0.5 720
EXTRA=(--roi 256 767 256 767 258 769); run "1024^2 x 720 -> central 512^3 ROI of the natural grid" 1024 1024 0.2 0.2 0 0 500 500 0.5 720
EXTRA=(--vol 512 512 64 0.0998692); run "512^2 x 1440 -> 512x512x64 slab" 512 512 0.2 0.2 0 0 500 500 0.25 1440
EXTRA=(--slabs 8); run "1024^2 x 360, natural volume in 8 slabs (8 tasks x 360)" 1024 1024 0.2 0.2 0 0 500 500 1.0 360
