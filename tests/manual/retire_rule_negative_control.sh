#!/bin/bash
# Negative control of tests/test_gpu_parity.py::test_retire_step_due_at_a_refused_hand_over: the same test against a build with
# the RETIRE rule as it was before commit 9b51a4a (-DKOMB_TEST_OLD_RETIRE_RULE) must FAIL (the host reports the skipped step).
# Run on the GPU box; prints "negative control ok" when the old rule is caught.
cd "$(dirname "$0")/../.."
make -s -j8 -C komb_amd/csrc OUT=../libv/oldretire EXTRA=-DKOMB_TEST_OLD_RETIRE_RULE ../libv/oldretire/libkomb_accel.so || exit 2
if KOMB_ACCEL_LIB=komb_amd/libv/oldretire/libkomb_accel.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k retire_step_due > gpurun_out/retire_negative_control.log 2>&1; then
    echo "negative control FAILED: the old rule passed the regression test"; tail -5 gpurun_out/retire_negative_control.log; exit 1
fi
grep -E "without a RETIRE step|AssertionError|Error" gpurun_out/retire_negative_control.log | head -3
echo "negative control ok: the old rule is caught"
