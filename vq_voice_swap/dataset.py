from vq_voice_swap_amd.audio import ChunkReader, ChunkWriter, decode_u_law, encode_u_law  # noqa: F401
