set -e
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp
rm -rf $O/seq_student
rocprofv3 --kernel-trace --output-format csv -d $O/seq_student -- python3 $R/bench.py --config student --steps 3 --warmup 2 --no-cpu-baseline --no-hbm-table > $O/seq_student.log 2>&1
python3 $R/tools/prof_sequence.py $O/seq_student adamw_multi_kernel > $O/seq_student_all.txt
python3 $R/tools/prof_sequence.py $O/seq_student adamw_multi_kernel gemm_bf16_nt_v7 gemm_bf16_nt_v6_kernel\<1 attn_fwd_dh64 layernorm_fwd_reg_kernel\<true > $O/seq_student_train.txt
rm -rf $O/seq_student
wc -l $O/seq_student_all.txt $O/seq_student_train.txt
