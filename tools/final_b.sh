bash tools/profile_round.sh r04 2>&1 | tail -3
