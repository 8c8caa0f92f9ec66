#!/bin/bash
O=gpurun_out/c11; mkdir -p $O
timeout 900 python -m pytest tests/test_raw28.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
timeout 100 python tools/raw28_probe.py 2>&1 | tail -1 | cut -c1-300
