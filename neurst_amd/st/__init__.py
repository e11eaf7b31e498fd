"""Host-side mirror of the reference's `st` package for the TransformerTPE hot path:
same module paths, class names and call signatures (st.models.TransformerTPE,
st.models.encoders.*, st.models.decoders.*, st.models.tools.utils, st.layers.*), with
bodies that launch the gfx950 kernels of libneurst_hip.so instead of building a
TensorFlow graph.  See INTEGRATION.md."""
