#!/bin/bash
# r04 GPU session 27: full GPU suite + smoke() on the final code of the round (after the batch / hybrid-map entries)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_27; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -8 | tee $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.log
