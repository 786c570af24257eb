# A/B of the second stream's priority (lowest, the default since round 4, against normal) in PARIS's loop through paris::hip:
# loop throughput and, from a kernel trace, how long a filter launch takes beside the fused kernel and how busy the fused launches are
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=$R/paris_amd/host/demo
for n in 2048 1024; do
  np=$([ $n = 2048 ] && echo 480 || echo 720)
  for prio in low normal low normal; do
    echo -n "n=$n $prio: "; PARIS_HIP_BP_STREAM_PRIORITY=$prio $D/paris_hip_demo $n $n 0.2 0.2 0 0 500 500 0.5 $np lcg /dev/null --cycle 48 --no-out | sed -n 2p | cut -c1-60
  done
  for prio in low normal; do
    ( cd /tmp && PARIS_HIP_BP_STREAM_PRIORITY=$prio rocprofv3 --kernel-trace --output-format csv -d /tmp/prio_$prio_$n -- $D/paris_hip_demo $n $n 0.2 0.2 0 0 500 500 0.5 $np lcg /dev/null --cycle 48 --no-out > /dev/null 2>&1 )
    echo "-- n=$n priority $prio, kernel trace:"; python $R/tools/timeline.py /tmp/prio_$prio_$n | grep -E "bp_fused_kernel|filter_rows|busy|gaps between"
    rm -rf /tmp/prio_$prio_$n
  done
done
