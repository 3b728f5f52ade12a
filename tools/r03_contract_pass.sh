#!/bin/bash
# round 3, contract 2 vs contract 1 on the GPU: exhaustive unary / pair checks, then kernel times per build (same tool as profiles/r02/variants*)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "match_the_contract or numerics_contract" -s 2>&1 | tail -8 > gpurun_out/contract_tests.log
python tools/variants.py portal_in_portal:3840:2160:40:1 triple_portal:3840:2160:40:1 monoportal:1920:1080:20:1 mobius_monoportal:3840:2160:64:1 \
   r3_all r3_all_w3 r3_all_w4 r3_v1_all r3_v1_all_w4 r3_dyn r3_v1_dyn r3_ints r3_v1_ints r3_fast_all_w4 > gpurun_out/variants1_contract.jsonl 2> gpurun_out/variants1_contract.err
tail -8 gpurun_out/contract_tests.log
cat gpurun_out/variants1_contract.jsonl
