#!/bin/bash
# round 4, second GPU call: probes with hashes (product vs race flavour must print the same hashes), the whole GPU suite on the
# product library, then the whole suite again on the race-provocation library (DBN_LIB_PATH)
O=gpurun_out/r04b; mkdir -p $O
tools/probes/repro 300 0 > $O/repro_plain.txt 2>&1
tools/probes/repro_race 300 1 > $O/repro_race_stress.txt 2>&1
diff <(grep hash $O/repro_plain.txt | sed 's/launches.*hash//') <(grep hash $O/repro_race_stress.txt | sed 's/launches.*hash//') && echo "product and race flavour: identical hashes"
python -m pytest tests -m gpu -q --timeout=1200 > $O/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -25 $O/gpu_tests.log
DBN_LIB_PATH=$PWD/db_text_minimal_amd/libdbnet_hip_race.so python -m pytest tests -m gpu -q --timeout=2400 -p no:cacheprovider > $O/gpu_tests_race.log 2>&1; echo "race pytest rc $?"; tail -25 $O/gpu_tests_race.log
