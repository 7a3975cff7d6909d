"""Reference-side binding stubs: the files a maintainer of zhangyp15/OccFormer drops into the reference tree to run its
unmodified Python over liboccformer_hip.so."""
