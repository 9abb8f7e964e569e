"""Stand-in for the nerfstudio package layout (tests only, see tests/stubs/README.md)."""
