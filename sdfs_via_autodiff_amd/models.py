"""
Model calibrations: host-side mirror of the reference's parameter classes.

SSY  -- code/ssy/ssy_model.py:50-81 (Schorfheide-Song-Yaron, 4 states)
GCY  -- code/gcy/gcy_model.py:43-75 (Gomez-Cram-Yaron, 6 states)

Same keyword names (Greek identifiers, as in the reference), same defaults and the
same ``params`` tuple order, so ``discretize_ssy(SSY(), shapes)`` reads like the
reference's drivers.
"""
import numpy as np


class SSY:
    """SSY parameters; ``params`` = (β, γ, ψ, μ_c, ρ, φ_z, φ_c, ρ_z, ρ_c, ρ_λ, s_z, s_c, s_λ)."""

    def __init__(self,
                 β=0.999, γ=8.89, ψ=1.97,
                 ρ=0.987, ρ_z=0.992, ρ_c=0.991, ρ_λ=0.959,
                 s_z=np.sqrt(0.0039), s_c=np.sqrt(0.0096), s_λ=0.0004,
                 μ_c=0.0016,
                 φ_z=0.215 * 0.0035 * np.sqrt(1 - 0.987**2),
                 φ_c=1.00 * 0.0035):
        self.β, self.γ, self.ψ = β, γ, ψ
        self.μ_c, self.φ_z, self.φ_c = μ_c, φ_z, φ_c
        self.ρ, self.ρ_z, self.ρ_c, self.ρ_λ = ρ, ρ_z, ρ_c, ρ_λ
        self.s_z, self.s_c, self.s_λ = s_z, s_c, s_λ
        self.θ = (1 - γ) / (1 - 1 / ψ)
        self.params = (β, γ, ψ, μ_c, ρ, φ_z, φ_c, ρ_z, ρ_c, ρ_λ, s_z, s_c, s_λ)


class GCY:
    """GCY parameters; ``params`` = (β, ψ, γ, ρ_λ, s_λ, μ_c, φ_c, ρ, ρ_π, φ_z, ρ_c, s_c,
    ρ_z, s_z, ρ_ππ, φ_zπ, ρ_zπ, s_zπ)."""

    def __init__(self,
                 β=0.9987, ψ=1.5, γ=13.01,
                 ρ_λ=0.981, s_λ=0.12 * 0.0015,
                 μ_c=0.0016, φ_c=0.0015,
                 ρ=0.983, ρ_π=-0.0075, φ_z=0.13 * 0.0015,
                 ρ_c=0.992, s_c=0.104,
                 ρ_z=0.980, s_z=0.09,
                 ρ_ππ=0.985, φ_zπ=0.08 * 0.0015,
                 ρ_zπ=0.970, s_zπ=0.271):
        self.β, self.ψ, self.γ = β, ψ, γ
        self.ρ_λ, self.s_λ, self.μ_c, self.φ_c, self.ρ = ρ_λ, s_λ, μ_c, φ_c, ρ
        self.ρ_π, self.φ_z, self.ρ_c = ρ_π, φ_z, ρ_c
        self.s_c, self.ρ_z, self.s_z = s_c, ρ_z, s_z
        self.ρ_ππ, self.φ_zπ, self.ρ_zπ, self.s_zπ = ρ_ππ, φ_zπ, ρ_zπ, s_zπ
        self.θ = (1 - γ) / (1 - 1 / ψ)
        self.params = (β, ψ, γ, ρ_λ, s_λ, μ_c, φ_c, ρ, ρ_π, φ_z, ρ_c, s_c, ρ_z, s_z,
                       ρ_ππ, φ_zπ, ρ_zπ, s_zπ)
