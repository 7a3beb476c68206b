#!/bin/bash
O=gpurun_out/r04f; mkdir -p $O
DBN_STAGGER=0 DBN_PATCH_F32=0 python tools/launch_table.py f32 --order > $O/lt_s0.txt 2>&1
DBN_STAGGER=1000 DBN_PATCH_F32=0 python tools/launch_table.py f32 --order > $O/lt_s1000.txt 2>&1
DBN_STAGGER=1000 DBN_PATCH_F32=1 python tools/launch_table.py f32 --order > $O/lt_s1000_p1.txt 2>&1
paste <(grep igemm $O/lt_s0.txt | cut -c1-12) <(grep igemm $O/lt_s1000.txt | cut -c1-130) | head -90
head -1 $O/lt_s0.txt $O/lt_s1000.txt $O/lt_s1000_p1.txt
