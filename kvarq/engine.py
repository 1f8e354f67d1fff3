"""``kvarq.engine`` -- KvarQ's scan engine on an AMD MI355X (replaces the C extension built from csrc/workhorse.c;
module contents as at workhorse.c:1567-1596: config, get_config, findseqs, stats, stop, test, Hit)."""
from kvarq_amd.engine import (config, get_config, findseqs, stats, stop, test, Hit,      # noqa: F401
                              install_sigint_counter)

# the C extension installs a counting SIGINT handler when it is imported (csrc/workhorse.c:133-136, 1632); the CLI's
# "press CTRL-C twice" logic (kvarq/cli.py:156-164) reads it back through stats()['sigints']
install_sigint_counter()
