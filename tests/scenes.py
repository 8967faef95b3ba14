"""Seeded synthetic scenes for the tests: thin re-export of the package's generator so tests and
bench.py build byte-identical inputs."""
from easy_gaussian_splatting_amd.synthetic import *  # noqa: F401,F403
from easy_gaussian_splatting_amd.synthetic import config_heavy, config_long_lists, config_bench_1m, config_s1, config_s2, config_s3, config_s5, dense_scene, look_at_circle, make_scene  # noqa: F401
