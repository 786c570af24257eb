# A/B of the workgroup -> tile order for a volume whose z tile count does not divide among the XCDs (1024 x 1024 x 1029: 33 tiles of 32
# slices for the fused kernel), through PARIS's per-projection loop (paris_hip_demo --order N): 18 = z tiles dealt + shared tail planes
# (round 4 default), 5 = a contiguous eighth per XCD (what rounds 1-3 fell back to), 17 / 15 = y tiles dealt in eights / pairs
D=paris_amd/host/demo
for r in 1 2; do for o in 18 5 17 15; do
  echo -n "order $o: "; $D/paris_hip_demo 1024 1024 0.2 0.2 0 0 500 500 0.5 720 lcg /dev/null --cycle 48 --no-out --order $o | sed -n 2p | cut -c1-60
done; done
for r in 1 2; do for o in 18 5 17; do
  echo -n "512^2 order $o: "; $D/paris_hip_demo 512 512 0.2 0.2 0 0 500 500 0.234375 1536 lcg /dev/null --cycle 48 --no-out --order $o | sed -n 2p | cut -c1-60
done; done
