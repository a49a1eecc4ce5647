#!/bin/bash
# second soak of the final library, other seeds
SEED=1200000 N_DEC=9000 N_COR=9000 N_ENC=20000 bash scripts/r3_soak.sh
