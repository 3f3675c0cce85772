"""Harness that lets the reference's own Python run in the build container (TEST INFRASTRUCTURE).

`shims.py` stands in for the three I/O-only third-party modules this image lacks.  Everything here needs
`/root/reference` and is therefore only used by generators and tests that skip when it is absent."""
