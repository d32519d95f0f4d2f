set -e
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp
rm -rf $O/seq_teacher
rocprofv3 --kernel-trace --output-format csv -d $O/seq_teacher -- python3 $R/bench.py --config teacher --steps 3 --warmup 2 --no-cpu-baseline --no-hbm-table > $O/seq_teacher.log 2>&1
python3 $R/tools/prof_sequence.py $O/seq_teacher adamw_multi_kernel > $O/seq_teacher_all.txt
python3 $R/tools/prof_sequence.py $O/seq_teacher adamw_multi_kernel gemm_bf16_nt_v7 gemm_bf16_nt_v6_kernel\<1 attn_fwd_dh64 layernorm_fwd_reg_kernel\<true > $O/seq_teacher_train.txt
rm -rf $O/seq_teacher
wc -l $O/seq_teacher_all.txt $O/seq_teacher_train.txt
