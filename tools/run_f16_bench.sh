set -e
mkdir -p gpurun_out
for w in hrnet_w32 hrnet_w48_384_udp_flip; do
  b=128; [ $w = hrnet_w48_384_udp_flip ] && b=64
  timeout -k 10 300 python bench.py --workload $w --batch $b --amp O2 --steps 10 --warmup 3 --no-cpu-baseline --layers gpurun_out/f16_layers_$w.csv > gpurun_out/f16_$w.json 2> gpurun_out/f16_$w.err
  python - <<PY
import json
r=json.load(open("gpurun_out/f16_$w.json"))
print("$w", r["value"], r["ms_per_step"], r["roofline"]["all_conv_launches"], r["roofline"]["kernel"], r["roofline"]["achieved"])
PY
done
