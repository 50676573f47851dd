"""promptcache_amd -- MI355X-native prompt-cache prefill path (drop-in for the reference's hot path)."""
