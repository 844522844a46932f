# GPU box: repeat the parity suite N times (flake hunt); prints one line per run plus any assertion text
N=${1:-5}
for i in $(seq 1 $N); do python -m pytest tests -m gpu -x -q 2>&1 | grep -E "AssertionError|rel err|passed|failed" | head -3; done
