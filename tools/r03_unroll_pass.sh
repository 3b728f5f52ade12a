mkdir -p gpurun_out/r03
python tools/variants.py portal_in_portal:3840:2160:40:1 tests/corpus/scenes/portal_in_portal_plus_ultra.ron:3840:2160:40:1 tests/corpus/scenes/recursive_space.ron:3840:2160:40:1 tests/corpus/scenes/matryoshka.ron:3840:2160:40:1 r3_all r3_all_nounroll r3_all_w4 r3_all_w4_nounroll r3_ints r3_ints_nounroll 2>/dev/null | tee gpurun_out/r03/variants5_unroll.jsonl
