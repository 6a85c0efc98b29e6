bash tools/profile_round.sh r03_h --steps 2 --warmup 1
