"""Fixture definitions shared by the CPU and GPU tests.

The *definitions* of the golden inputs (programs, seeds, dims) live in oracle/make_golden.py so
that the generator and the tests cannot drift; importing that module does not touch
/root/reference (only its ``main()`` does)."""
from oracle.make_golden import (  # noqa: F401
    LONG_CASES,
    SMALL_DIMS,
    VALIDITY_CASES,
    encode_programs,
    full_module_inputs,
    namespaces,
    small_dims,
    small_network_inputs,
)
