"""tehmm_amd: MI355X-native implementation of teHmm's hot path (emission log-likelihood,
forward / backward / Viterbi, Baum-Welch E-step) behind the reference's Python model API."""
__all__ = ["_lib", "_hmm", "_emission", "common", "track", "synth"]
