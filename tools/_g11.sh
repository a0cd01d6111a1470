run() { name=$1; shift; echo "== $name"; env "$@" python tools/step_profile.py --batch 256 --reps 2 2>&1 | grep isolated; }
run base X=1
run fat512 MOCR_DEC_FAT_ROWS=512
run fat512_b300 MOCR_DEC_FAT_ROWS=512 MOCR_DEC_BLOCKS=300
run b80 MOCR_DEC_BLOCKS=80
run b300 MOCR_DEC_BLOCKS=300
run latent256 MOCR_CLASSIC_ROWS=128
run latent256_fat512 MOCR_CLASSIC_ROWS=128 MOCR_DEC_FAT_ROWS=512
echo "== 64 rows"; python tools/step_profile.py --batch 64 --reps 2 2>&1 | grep isolated
echo "== 64 rows b300"; MOCR_DEC_BLOCKS=300 python tools/step_profile.py --batch 64 --reps 2 2>&1 | grep isolated
echo "== 64 rows b80"; MOCR_DEC_BLOCKS=80 python tools/step_profile.py --batch 64 --reps 2 2>&1 | grep isolated
