python tools/exp_run.py a_pk d_nopk n_feat n_feat_stage 2>&1 | tail -4
for n in d_nopk n_feat n_feat_stage; do python tools/step_cycles.py $n brief 2>&1 | grep -v amdgpu.ids | tail -1; done
