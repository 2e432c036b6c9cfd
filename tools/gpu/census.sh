#!/bin/bash
cd $GRAFT_REPO_ROOT
N=${N:-2048} STEPS=${STEPS:-5,20,35} python tools/diag/leap_contact_census.py 2>&1 | tail -60
