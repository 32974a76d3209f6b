cd $GRAFT_REPO_ROOT
for i in $(seq 1 30); do timeout 120 python -m pytest tests/test_gpu_models.py -x -q -k "test_failed_capture_falls_back" > /tmp/fl.log 2>&1; if grep -q "1 failed" /tmp/fl.log; then echo "RUN $i FAILED"; grep -E "^E |assert|Error" /tmp/fl.log | head -20; fi; done; echo done
