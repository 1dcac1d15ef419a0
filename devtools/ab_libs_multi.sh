#!/bin/bash
# several workloads under each of several whole libraries: bash devtools/ab_libs_multi.sh "name name" "args;args;..."
IFS=';' read -ra WL <<< "$2"
for w in "${WL[@]}"; do bash devtools/ab_libs.sh "$1" "$w"; done
