C=tdrn_amd/csrc
O=gpurun_out/conv_check_8.txt
timeout 300 $C/_build/conv_check > $O 2>&1; echo rc=$? >> $O
grep -v "differ" $O | tail -40; grep -c differ $O
