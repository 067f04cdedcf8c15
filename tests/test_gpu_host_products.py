"""The host-side (numpy) product functions whose parity is pinned to reference goldens -- STNO builder F1, STNO seek windows
F12, long-form segment retrieval F16, window-relative timestamp folding F17, the library's symbol export -- are CPU tests
(tests/test_host_logic.py); this module re-runs the same checks in the `-m gpu` set, so that the GPU box's report covers them
with the interpreter / numpy / library build that actually runs the kernels there."""
import pytest

from tests import test_host_logic as H

pytestmark = pytest.mark.gpu


def test_f1_stno_builder_bit_exact_on_the_gpu_box():
    H.test_product_stno_builder_bit_exact_vs_reference_golden()
    H.test_collate_stno_padding_is_silence()


def test_f12_seek_windows_on_the_gpu_box():
    H.test_product_stno_seek_windows_match_reference_golden()


def test_f16_retrieve_segment_on_the_gpu_box():
    H.test_product_retrieve_segment_matches_reference_golden()


def test_f17_fix_timestamps_on_the_gpu_box():
    H.test_product_fix_timestamps_matches_reference_golden_and_oracle()


def test_library_symbols_and_no_cpu_fallback_on_the_gpu_box():
    H.test_library_exports_every_declared_symbol()
    H.test_no_cpu_fallback_and_oracle_not_imported_by_product()


def test_f18_temperature_fallback_decisions_on_the_gpu_box():
    H.test_product_temperature_fallback_matches_transformers_golden()
