#!/bin/bash
set -o pipefail
bash tools/run_final.sh
