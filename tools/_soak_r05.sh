bash tools/soak_fuzz.sh r05_final 1 3000 300 300
echo "== split fuzz 150 seeds"; MM_FUZZ_SPLIT=150 timeout 1500 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "fuzz_split" 2>&1 | tail -3
echo "== two-rank tests x 8"; for i in 1 2 3 4 5 6 7 8; do timeout 600 python -m pytest tests/test_gpu_multi.py -x -q -m gpu -k "two_ranks_one_gpu or two_contexts_one_gpu or several_contexts" 2>&1 | tail -1; done
