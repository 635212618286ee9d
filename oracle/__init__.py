"""Test infrastructure only (CPU oracle + golden-vector generator).  Never imported by fastspeech2_amd/."""
