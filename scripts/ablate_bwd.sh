# kernel times of the C3 update pass for the ablation builds of wide_fused_bwd_kernel (build: for n in 1 2 3 4; do bash scripts/build_variant.sh ablb$n -DCRL_ABL_B=$n wide; done)
for v in default ablb1 ablb2 ablb3 ablb4; do bash scripts/c3_kernels.sh $v --no-extras 2>&1 | grep -E "==|fused_bwd|fused_fwd|wgrad_gen"; done
