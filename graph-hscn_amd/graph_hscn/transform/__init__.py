from .posenc import compute_posenc_stats, eigvec_normalizer, get_lap_decomp_stats

__all__ = ["compute_posenc_stats", "eigvec_normalizer", "get_lap_decomp_stats"]
