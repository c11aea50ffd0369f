#!/bin/bash
# Round 6: the full-size parity gate again (tools/full_parity.py: EVERY read of a BASELINE config against the oracle, all triples +
# counts) on the round's final tree -- the LDS form, the table form pinned, and cfg 3 / cfg 2 with 1 % of reads carrying a '.' and
# 1 % an ambiguity code or a byte of no meaning (the first pass's recode, the second pass's list).   usage: tools/r06_full_parity.sh > out.jsonl
cd "$(dirname "$0")/.."
run() { echo "# $*" ; env $ENVS python tools/full_parity.py "$@" 2>/dev/null | tail -1; }
ENVS="" run --config 3
ENVS="" run --config 3 --table
ENVS="" run --config 2
ENVS="" run --config 4
ENVS="" run --config 5
ENVS="" run --config 1
ENVS="FQTK_SYNTH_PDOT=0.00063 FQTK_SYNTH_PIUPAC=0.00063" run --config 3
ENVS="FQTK_SYNTH_PDOT=0.00063 FQTK_SYNTH_PIUPAC=0.00063" run --config 3 --table
ENVS="FQTK_SYNTH_PDOT=0.0066 FQTK_SYNTH_PIUPAC=0.0066" run --config 3
ENVS="FQTK_SYNTH_PDOT=0.00125 FQTK_SYNTH_PIUPAC=0.00125" run --config 2
ENVS="FQTK_SYNTH_PDOT=0.001 FQTK_SYNTH_PIUPAC=0.001" run --config 5
# tables the perfect-hash LDS form serves (round 6): 384 x 24 (a 12+12 dual index), 440 x 24, 330 x 30 -- clean, and with odd bytes in the reads
ENVS="" run --custom 384,24,100000000
ENVS="" run --custom 384,24,100000000 --table
ENVS="FQTK_SYNTH_PDOT=0.0005 FQTK_SYNTH_PIUPAC=0.0005" run --custom 384,24,100000000
ENVS="" run --custom 440,24,50000000
ENVS="FQTK_SYNTH_PDOT=0.0005 FQTK_SYNTH_PIUPAC=0.0005" run --custom 330,30,50000000
