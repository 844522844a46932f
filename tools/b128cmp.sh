for sm in 1 0 1 0; do for b in 128 1 32 256; do
  printf "small=%s B=%s " $sm $b; SYNERGY_HIP_EARLY_RM=$((1023 + 1024 * sm)) python bench.py --lmk-only --batch $b --steps 200 --warmup 20 --overlap 0 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done; done
