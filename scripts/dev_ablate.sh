C=tdrn_amd/csrc
O=gpurun_out/conv_ablate_1.txt
: > $O
for A in 1 2 4 8 16; do
  echo "== ablate $A" >> $O
  timeout 60 $C/_build_a$A/conv_check 32 80 80 256 256 0 1 1 20 2>&1 | tail -1 >> $O
  timeout 60 $C/_build_a$A/conv_check 32 40 40 512 512 0 1 1 20 2>&1 | tail -1 >> $O
done
cat $O
