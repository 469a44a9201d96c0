for n in a_pk h_stage t_nopark t_noparklo t_nofragf t_nofragr t_nolold t_nodma t_nosums t_novmemf t_all; do python tools/step_cycles.py $n brief 2>&1 | grep -v amdgpu.ids | tail -1; done
