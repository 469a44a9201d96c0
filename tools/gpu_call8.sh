python tools/exp_run.py d_nopk n_feat n_ril2 n_ril3 n_ril3_st n_ril4 2>&1 | tail -6
for n in n_ril3 n_ril3_st; do python tools/step_cycles.py $n brief 2>&1 | grep -v amdgpu.ids | tail -1; done
