# round 6, call 3: ablations of the producer / consumer DCN kernel (timing only; the variants compute wrong results by construction)
for sig in 1.5 0.3; do
export OFFSET_SIGMA=$sig
echo "#### OFFSET_SIGMA=$sig"
for lib in hip pc_nofar pc_nogather pc_noblend pc_nowrite pc_nomfma pc_nodma pc_noaread pc_prodonly pc_consonly; do
  for shape in "152 272 64 64 16 64" "76 136 128 64 16 64"; do
    echo -n "$lib: "; DEFT_HIP_LIB=$PWD/deft_amd/lib/libdeft_$lib.so timeout 120 python tools/probe/dcnp_one.py $shape 20 2>&1 | tail -1
  done
done
echo -n "one-role kernel: "; DEFT_DCN_PC=0 timeout 120 python tools/probe/dcnp_one.py 152 272 64 64 16 64 20 2>&1 | tail -1
done
