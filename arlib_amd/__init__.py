"""arlib_amd -- MI355X-native implementation of ARLib's embedding-recommender training and white-box
attack hot path.  The compute lives in arlib_amd/lib/libarlib_amd.so (HIP, gfx950);
this package is the host-side mirror of the reference's Python interface for that path."""
__version__ = '0.1.0'
