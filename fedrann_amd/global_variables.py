"""Process-wide settings, as in the reference's fedrann/global_variables.py:4-7."""
threads: int = 1
seed: int = 5349875
output_dir: str = ""
temp_dir: str = ""
