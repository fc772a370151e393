"""phamers_amd -- MI355X-native k-mer count + phage-score path behind the
PhaMers function signatures."""
__version__ = "0.1.0"
