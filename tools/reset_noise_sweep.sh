for s in 0.0 0.33 1.0; do
  timeout -k 10 300 python tools/train_reset_noise.py --scale $s --epochs 1200 --seed 42 2>&1 | grep "^epoch" | awk 'NR%100==0 || 0' | awk '{print $2, $NF}' | tr '\n' ' ' > gpurun_out/rn_$s.txt || exit 1
  echo >> gpurun_out/rn_$s.txt
done
