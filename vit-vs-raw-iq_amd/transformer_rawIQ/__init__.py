"""Import-path mirror of the reference's Transformer_Thesis/transformer_rawIQ package."""
