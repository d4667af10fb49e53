cp bgflow_amd/libbgflow_amd.so /tmp/lib_orig.so
for f in gpurun_variants/lib_hw*.so; do
  cp $f bgflow_amd/libbgflow_amd.so
  echo "== $(basename $f) (exp log div)"
  python tools/accuracy_report.py 2>/dev/null | grep f16x2
  python bench.py --no-cpu-baseline --no-extras --kl-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['value'], d['ms_per_step'])"
done
cp /tmp/lib_orig.so bgflow_amd/libbgflow_amd.so
