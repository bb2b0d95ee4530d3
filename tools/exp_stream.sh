mkdir -p gpurun_out/r4_g8
P=tools/microbench/spin_probe
O=gpurun_out/r4_g8/spin_probe.txt
for args in "248 1024 0 2 0" "200 1024 0 2 0" "256 1024 0 2 0" "248 1024 0 2 1" "128 1024 1 2 0"; do
  timeout -k 5 30 $P $args >> $O 2>&1
done
echo done
