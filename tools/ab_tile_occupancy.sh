# same-device A/B of the single-projection kernel's slices in flight (--unroll) and workgroups per CU (capped by the LDS size: 24 KiB -> as many
# as the registers allow, 36-40 KiB -> 4, 53 KiB -> 3) per configuration
C="--cpu-budget 0 --cpu-c1 0 --live-traffic 0 --workloads 0 --paris-loop 0 --fused-steps 0"
python tools/ab_args.py --rounds 2 --common "$C --workload c3 --slices 256 --steps 10 --warmup 2" "" "--lds-bytes 40960" "--unroll 2" "--unroll 2 --lds-bytes 40960" "--tz 16 --unroll 1 --lds-bytes 40960" "--lds-bytes 32768"
python tools/ab_args.py --rounds 2 --common "$C --workload c2 --steps 10 --warmup 2" "" "--lds-bytes 40960" "--unroll 2" "--tz 16 --unroll 1 --lds-bytes 40960" "--lds-bytes 32768"
python tools/ab_args.py --rounds 2 --common "$C --workload c5 --steps 4 --warmup 1 --batch 36 --spread 1" "" "--unroll 1 --lds-bytes 40960" "--unroll 1"
python tools/ab_args.py --rounds 2 --common "$C --workload c1 --steps 10 --warmup 2" "" "--lds-bytes 40960" "--unroll 1" "--unroll 1 --lds-bytes 40960"
