"""engineer.core of the reference: curve / shape initialisers and the 2-D feature-line loss."""
