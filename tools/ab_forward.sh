for i in 1 2; do
for v in base new; do
  if [ $v = base ]; then export FHELIN_LIB=$GRAFT_REPO_ROOT/tmp_variants/base/libfhelin_amd.so; else unset FHELIN_LIB; fi
  python bench.py --steps 6 --warmup 2 --no-ops --inflight 0 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'])"
done; done
