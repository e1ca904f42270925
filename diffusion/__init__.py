"""Drop-in alias of the reference package layout (diffusion/)."""
