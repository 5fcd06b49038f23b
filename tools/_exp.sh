cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_training.py tests/test_hip_cc_training.py -m gpu -x -q 2>&1 | tail -5
timeout 300 python tools/train_time.py 20 train_attn_split=0 2>&1 | tail -2
timeout 300 python tools/train_time.py 20 train_attn_split=1 2>&1 | tail -2
