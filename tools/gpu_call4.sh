python tools/exp_run.py a_pk h_stage 2>&1 | tail -3
python tools/step_cycles.py a_pk 2>&1 | grep -A1 "^default"
python tools/step_cycles.py h_stage 2>&1 | grep -A1 "^default"
