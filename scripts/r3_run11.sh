#!/bin/bash
# soak of the final library: random streams, damaged streams, encoder groups
SEED=1100000 N_DEC=9000 N_COR=9000 N_ENC=3000 bash scripts/r3_soak.sh
