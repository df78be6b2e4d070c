cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 600 python -m pytest tests -m gpu -x -q -k "gram or trace" 2>&1 | tail -2
timeout 300 python tools/lab/gram_check.py 2>&1 | grep shape | cut -c1-250
