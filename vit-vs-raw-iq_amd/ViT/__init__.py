"""Import-path mirror of the reference's Transformer_Thesis/ViT package."""
