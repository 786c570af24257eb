# A/B of PARIS_HIP_FILTER_DEFERRAL in the C++ mirror. Round 4 (profiles/r04_ab_filter_deferral_in_the_mirror.txt): 0 = paris_hip_demo, then the
# default, against 1 = paris_hip_demo_filter_deferral. Since round 5 the default is 2 (held back only where the filter can run in place, by
# reference) and paris_hip_demo_filter_at_once is the 0 build: tools/mirror_split.sh compares those. PARIS's
# per-projection loop, whole circles: natural volumes of 512^2 / 1024^2 / 2048^2 detectors, the 2048 x 2048 x 256 slab a rank of the 8-GPU
# configuration owns (2048^2 frames), and BASELINE config 1 (256^3 from 512^2 frames)
D=paris_amd/host/demo
run() { echo -n "$1 | $2: "; shift 2; "$@" | sed -n 2,3p | tr "\n" " " | sed -e 's/through paris.*of which://' | cut -c1-200; echo; }
for r in 1 2; do
  for v in demo demo_filter_deferral; do
    run "512^2 x 360 natural" $v $D/paris_hip_$v 512 512 0.2 0.2 0 0 500 500 1.0 360 lcg /dev/null --cycle 48 --no-out
    run "512^2 x 1536 natural" $v $D/paris_hip_$v 512 512 0.2 0.2 0 0 500 500 0.234375 1536 lcg /dev/null --cycle 48 --no-out
    run "1024^2 x 720 natural" $v $D/paris_hip_$v 1024 1024 0.2 0.2 0 0 500 500 0.5 720 lcg /dev/null --cycle 48 --no-out
    run "2048^2 x 480 natural" $v $D/paris_hip_$v 2048 2048 0.2 0.2 0 0 500 500 0.75 480 lcg /dev/null --cycle 48 --no-out
    run "2048^2 x 1440 -> 2048x2048x256 (config 4 slab)" $v $D/paris_hip_$v 2048 2048 0.2 0.2 0 0 500 500 0.25 1440 lcg /dev/null --cycle 48 --no-out --vol 2048 2048 256 0.0979666
    run "512^2 x 360 -> 256^3 (config 1)" $v $D/paris_hip_$v 512 512 0.2 0.2 0 0 500 500 1.0 360 lcg /dev/null --cycle 48 --no-out --vol 256 256 256 0.19973837
  done
done
