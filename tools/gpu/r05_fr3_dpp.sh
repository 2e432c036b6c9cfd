#!/bin/bash
# GPU box: fr3 kernel with the Cholesky in registers (DPP row broadcasts) against the LDS row form: recorded inputs A/B, then the fr3 tests with margins
cd $GRAFT_REPO_ROOT
bash tools/gpu/r05_ab_fr3.sh r5g ${VARIANTS:-v6lds product v6lds product}
rm -f gpurun_out/test_margins.jsonl
JUDO_RECORD_MARGINS=1 timeout 900 python -m pytest tests/test_gpu_fr3.py tests/test_gpu_edges.py -m gpu -q -W error::RuntimeWarning -p no:cacheprovider > gpurun_out/r5g/pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r5g/pytest.txt
tail -n 8 gpurun_out/r5g/pytest.txt | cut -c1-250
cp gpurun_out/test_margins.jsonl gpurun_out/r5g/
