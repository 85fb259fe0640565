for tw in 2048 3072 4096 5120 8192; do echo "target $tw"; BLASR_HIP_KBAND_TARGET_WAVES=$tw python tools/kband_bench.py --reads 1024 --len 30000 --k 2700 --repeat 1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in d['runs'][-1]['kernels']: print(k['name'], round(k['ms'],1), round(k['gcells_per_s'] or 0))
"; done
for k in 60 100 200 300 543; do for wc in 3 100000; do echo "k $k widecols $wc"; BLASR_HIP_KBAND_WIDE_COLS=$wc python tools/kband_bench.py --reads 4096 --len 10000 --k $k --repeat 1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in d['runs'][-1]['kernels']: print(k['name'], round(k['ms'],1), round(k['gcells_per_s'] or 0))
"; done; done
