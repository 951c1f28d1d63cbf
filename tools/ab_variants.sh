for lib in libgsamd.so libgsamd_xnocls.so libgsamd_xpb0.so libgsamd_xnobk.so libgsamd_xall3.so; do
  GS_LIB_PATH=$PWD/guidescan-cli_amd/$lib python bench.py --cpu-sample 0 --steps 5 --warmup 1 > gpurun_out/ab.json 2> gpurun_out/ab.err
  python3 -c "
import json
j=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('$lib', round(j['value']), round(j['ms_per_step'],2), j['detail']['k_search_ms_per_step'])"
done
