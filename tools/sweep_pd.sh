for lib in guidescan-cli_amd/libgsamd.so guidescan-cli_amd/libgsamd_pd7.so guidescan-cli_amd/libgsamd_pd6.so; do
  for cfg in "3 1000000" "4 200000" "5 100000"; do set -- $cfg
    GS_LIB_PATH=$PWD/$lib timeout -k 10 300 python bench.py --mismatches $1 --batch $2 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/w.json 2> gpurun_out/w.err
    python3 -c "
import json
j=json.loads(open('gpurun_out/w.json').read().strip().splitlines()[-1]); print('$lib', $1, round(j['value']), round(j['ms_per_step'],2), j['detail']['k_search_ms_per_step'], j['detail']['hits_per_guide'])"
  done
done
GS_NO_SPEC=1 python bench.py --cpu-sample 0 --steps 3 --warmup 1 > gpurun_out/w.json 2>/dev/null; python3 -c "
import json
j=json.loads(open('gpurun_out/w.json').read().strip().splitlines()[-1]); print('nospec', round(j['value']), round(j['ms_per_step'],2), j['detail']['k_search_ms_per_step'])"
