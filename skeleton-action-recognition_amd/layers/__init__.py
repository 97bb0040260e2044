"""Drop-in for the reference's `layers` package: `from layers.virtual_radar import VirtualRadar`."""
