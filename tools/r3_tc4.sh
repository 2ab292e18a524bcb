cd "$GRAFT_REPO_ROOT"
for L in "3x3 128->256 @52" "1x1 256->128 @52" "3x3 256->512 @26"; do
timeout -k 10 200 python tools/conv_bench.py --reps 20 --layers "$L" --variants "flat=0;tc=4;tc=4,stg=3;flat=1;flat=0" 2>&1 | grep -v amdgpu.ids
done
