"""
kvarq_amd -- MI355X-native implementation of KvarQ's read-scanning hot path
(the ``kvarq.engine`` module of the reference, csrc/workhorse.c) behind the
reference's own Python API.  See DESIGN.md and INTEGRATION.md.
"""
VERSION = '0.1.0'
