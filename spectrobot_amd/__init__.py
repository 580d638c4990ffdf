"""spectrobot_amd -- MI355X-native SpectRobot spectral hot path."""
